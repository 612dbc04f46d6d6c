// `up.forward` of the decoder (reference models/unet.py:50-59) as ONE kernel in inference:
//     y = relu(bn(conv3x3(cat([skip, ConvTranspose2d(x, k 2, s 2) + b]))))            (the first conv of `up.conv`)
//
// Why: at batch 16 the three ConvTranspose launches per stream take 0.3 ms at 1.4-3 TB/s, and their outputs
// (67-268 MB) are written once and read once more by the 3x3 conv that follows.  Both operators are linear, so the
// up half of the 3x3 conv is composed with the transposed conv ONCE, at weight-packing time
// (ammc_pack_up_conv_f32): output pixel (2 y + py, 2 x + px) sees the 2 x 2 source pixels (y + sy + py - 1,
// x + sx + px - 1) of x through filters that depend on its parity class (py, px) only,
//     W'[py][px][sy][sx][co][ci] = sum over the taps (r, s) whose up-pixel has that source pixel of
//                                  W3[co][c + cu][r][s] * Wt[ci][cu][dy][dx]       (summed over cu, in double),
// 8 c instead of 9 c multiply-adds per output and filter, no intermediate tensor, no ConvTranspose launch.  Zero padding
// composes exactly (a tap outside the image reads a source pixel outside the image: the halo, zero); the transposed
// conv's bias passes through the taps that are inside the image, i.e. it becomes a shift that depends on the border
// class of the output pixel: shift9[3][3][n] = shift + scale * (sum of the inside taps' W3 . b).
//
// Structure: the halo-patch kernel (conv_tap_s16.hip, 16x16x32 MFMA form) with the output pixels of a workgroup's
// 8 x 32 patch owned by PARITY CLASS: wave (py, px) holds the 4 x 16 pixels (2 y' + py, 2 x' + px) as four 16-pixel
// MFMA tiles.  Phase A = the skip half: ordinary 3x3 taps over the skip patch, whose columns are stored even / odd
// de-interleaved in LDS so that a tile's 16 stride-2 pixels are 16 consecutive LDS rows (conflict-free fragment reads
// as in the tap kernel); one filter slice per tap, shared by the four waves.  Phase B = the composed half: per
// 32-channel block of x a 6 x 18 source patch, per 2x2 tap one stage of four class slices (64 filters each), every wave
// reading ITS class - the same fragment reads per MFMA as phase A.  The LDS regions of the two phases overlay:
// 80 KB per workgroup, two workgroups per CU.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <stdio.h>

namespace ammc_s16 {

typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));

struct UpArgs {
  AmmcConvDesc d;        // the 3x3 conv over the SKIP tensor: x (halo corner), cin = c, w = full filter [n][kpad], scale, y ...
  const float* x2;       // the tensor the transposed conv reads (half resolution, halo 1, S16): its halo corner
  int64_t x2_bs, x2_rs, x2_ps;
  const float* w2;       // composed filters, S16 [n][kpad2], k = (tap2 * 4 + class) * cin2 + ch
  const float* shift9;   // [3][3][n]
  int cin2, ncc1, ncc2, ncc_full, kpad, kpad2, tiles_x, tiles_y, n_tiles;
};

constexpr float U_LO_SCALE = 2048.f;
constexpr float U_LO_INV = 1.f / 2048.f;
constexpr int U_TH = 8, U_TW = 32, U_HW = U_TW + 2;
constexpr int U_NT = 256;
constexpr int U_APIECES = (U_TH + 2) * U_HW * 8;                       // 2720 16-byte pieces of the skip patch
constexpr int U_AROUNDS = (U_APIECES + U_NT - 1) / U_NT;               // 11
constexpr int U_ASTAGE = U_AROUNDS * U_NT * 4;                         // floats (45056 B)
constexpr int U_SH = U_TH / 2 + 2, U_SW = U_TW / 2 + 2;                // 6 x 18 source pixels
constexpr int U_SPIECES = U_SH * U_SW * 8;                             // 864
constexpr int U_SROUNDS = (U_SPIECES + U_NT - 1) / U_NT;               // 4
constexpr int U_SSTAGE = U_SROUNDS * U_NT * 4;                         // floats (16 KB)
constexpr int U_B2SLOT = 4 * 32 * 32;                                  // ring slot: four class slices of 32 filters (16 KB)
constexpr int U_B2RING = 4;                                            // slots: three stages in flight behind the one in use
constexpr int U_B2J = 4;                                               // DMA instructions per lane per stage (32 rows x 8 pieces)

template <int TN>      // BN = 32 TN filters per workgroup: 64 (TN 2) or 128 (TN 4)
__global__ __launch_bounds__(U_NT, 2) void conv_up_s16_kernel(UpArgs a) {
  constexpr int BN = 32 * TN;
  constexpr int FT = 2 * TN;                   // 16-filter tiles
  constexpr int FC = 2;                        // filter tiles whose fragments are live at once
  constexpr int BJ = BN * 8 / U_NT;            // phase A: filter pieces per thread per tap
  constexpr int B_STAGE = BN * 32;             // floats
  constexpr int SPT = BN / 32;                 // phase B: stages (32 filters) per 2x2 tap
  constexpr int SPC = 4 * SPT;                 //          stages per 32-channel block of x
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                            // phase A: skip patch, then two filter stages
  float* Bs = smem + U_ASTAGE;
  float* Ss = smem;                            // phase B (overlays phase A): source patch, then two stages of class slices
  float* B2 = smem + U_SSTAGE;
  static_assert(U_ASTAGE + 2 * B_STAGE <= U_SSTAGE + U_B2RING * U_B2SLOT, "phase A fits the phase B footprint");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;                   // = parity class of this wave's output pixels
  const int py = wave >> 1, px = wave & 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  const AmmcConvDesc& d = a.d;

  const int logical = ammc_xcd_remap(blockIdx.x, gridDim.x);
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);     // see conv_tap_s16.hip: de-phase the two workgroups of a CU
  const int n0 = (logical % a.n_tiles) * BN;
  int sp = logical / a.n_tiles;
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int b = sp / a.tiles_y;
  const int y0 = ty * U_TH, x0 = tx * U_TW;

  const float* xpatch = d.x + ((int64_t)b * d.x_bs + (int64_t)y0 * d.x_rs + (int64_t)x0 * d.x_ps);
  const float* spatch = a.x2 + ((int64_t)b * a.x2_bs + (int64_t)(y0 >> 1) * a.x2_rs + (int64_t)(x0 >> 1) * a.x2_ps);

  // DMA pieces: piece p -> LDS row R = p >> 3, physical slot p & 7, which holds source piece pi((p & 7) ^ (R & 7))
  // (pi swaps 2 <-> 3 and 6 <-> 7; conv_tap_s16.hip, MF = 1).  Skip patch: LDS row R = hy * 34 + pos holds column
  // hx = 2 pos (pos < 17) or 2 (pos - 17) + 1: even columns first, then the odd ones.
  int a_off[U_AROUNDS];
#pragma unroll
  for (int j = 0; j < U_AROUNDS; ++j) {
    int p = j * U_NT + tid;
    p = p < U_APIECES ? p : U_APIECES - 1;
    const int R = p >> 3;
    int ls = (p & 7) ^ (R & 7);
    ls ^= (ls >> 1) & 1;
    const int hy = R / U_HW, pos = R - hy * U_HW;
    const int hx = pos < 17 ? 2 * pos : 2 * (pos - 17) + 1;
    a_off[j] = (int)((int64_t)hy * d.x_rs + (int64_t)hx * d.x_ps) + 4 * ls;
  }
  int s_off[U_SROUNDS];
#pragma unroll
  for (int j = 0; j < U_SROUNDS; ++j) {
    int p = j * U_NT + tid;
    p = p < U_SPIECES ? p : U_SPIECES - 1;
    const int R = p >> 3;
    int ls = (p & 7) ^ (R & 7);
    ls ^= (ls >> 1) & 1;
    const int sy = R / U_SW, sx = R - sy * U_SW;
    s_off[j] = (int)((int64_t)sy * a.x2_rs + (int64_t)sx * a.x2_ps) + 4 * ls;
  }
  // filter rows of a stage are stored tile by tile: LDS row 16 t + r holds filter 32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3)
  int sl = (tid & 7) ^ ((tid >> 3) & 7);
  sl ^= (sl >> 1) & 1;
  // (thread tid serves LDS rows 32 j + (tid >> 3): the permutation stays inside a block of 32 rows, so round j is a
  // uniform offset from round 0 - one address register instead of BJ)
  int row0 = tid >> 3;
  row0 = (((row0 >> 2) & 3) << 3) | (((row0 >> 4) & 1) << 2) | (row0 & 3);
  const float* b_src0 = d.w + (int64_t)(n0 + row0) * a.kpad + 4 * sl;
  // phase B stage: every wave streams ITS class slice (32 filter rows x 8 pieces = 4 per lane) and nobody else reads it:
  // the slices need no workgroup barrier, only the issuing wave's own COUNTED vmcnt (a ring of four slots, three stages
  // in flight).  Piece p = j * 64 + lane -> row p >> 3 = 8 j + (lane >> 3), after the tile-order permutation
  // 16 (j & 1) + 4 (j >> 1) + a lane part
  const int sl2 = ((lane & 7) ^ ((lane >> 3) & 7)) ^ ((((lane & 7) ^ ((lane >> 3) & 7)) >> 1) & 1);
  const int b2_lane = ((((lane >> 3) >> 2) & 1) * 8 + ((lane >> 3) & 3)) * a.kpad2 + wave * a.cin2 + 4 * sl2;
  const float* w2base = a.w2 + (int64_t)n0 * a.kpad2;

#define UP_ISSUE_A(cc)                                                                \
  _Pragma("unroll") for (int j_ = 0; j_ < U_AROUNDS; ++j_) {                          \
    const float* src_ = xpatch + a_off[j_] + (cc) * 32;                               \
    float* dst_ = As + (j_ * U_NT + wave * 64) * 4;                                   \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }
#define UP_ISSUE_S(cb)                                                                \
  _Pragma("unroll") for (int j_ = 0; j_ < U_SROUNDS; ++j_) {                          \
    const float* src_ = spatch + s_off[j_] + (cb) * 32;                               \
    float* dst_ = Ss + (j_ * U_NT + wave * 64) * 4;                                   \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }
#define UP_ISSUE_B(chunk, stage)                                                      \
  _Pragma("unroll") for (int j_ = 0; j_ < BJ; ++j_) {                                 \
    const float* src_ = b_src0 + (int64_t)j_ * 32 * a.kpad + (chunk) * 32;            \
    float* dst_ = Bs + (stage) * B_STAGE + (j_ * U_NT + wave * 64) * 4;               \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }
#define UP_ISSUE_B2(tap2, fblk, cb, slot)                                             \
  _Pragma("unroll") for (int j_ = 0; j_ < U_B2J; ++j_) {                              \
    const float* src_ = w2base + (int64_t)((fblk) * 32 + 16 * (j_ & 1) + 4 * (j_ >> 1)) * a.kpad2 + b2_lane + \
                        (tap2) * 4 * a.cin2 + (cb) * 32;                              \
    float* dst_ = B2 + (slot) * U_B2SLOT + wave * (32 * 32) + j_ * 64 * 4;            \
    __builtin_amdgcn_global_load_lds(src_, dst_, 16, 0, 0);                           \
  }
#define UP_WAIT_ALL()                                  \
  {                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  }
#define UP_WAIT_OWN(n)                                                               \
  {                                                                                  \
    if ((n) >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   \
    else if ((n) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                               \
  }

  f32x4 acc[4][FT];                           // [pixel tile y'][filter tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int slot_hi = 2 * g4 + (g4 & 1), slot_lo = slot_hi ^ 1;
  const int swzb = l15 & 7;

  // one MFMA group: the four pixel tiles (fragments ah / ax / al2 resident) against filter tiles [f0, f0 + FC) read at Bc
#define UP_MFMAS(Bc, f0)                                                                                   \
  {                                                                                                        \
    f16x8u bh[FC], bl[FC];                                                                                 \
    _Pragma("unroll") for (int j = 0; j < FC; ++j) {                                                       \
      bh[j] = *reinterpret_cast<const f16x8u*>((Bc) + (j * 16 + l15) * 32 + ((slot_hi ^ swzb) << 2));      \
      bl[j] = *reinterpret_cast<const f16x8u*>((Bc) + (j * 16 + l15) * 32 + ((slot_lo ^ swzb) << 2));      \
    }                                                                                                      \
    _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)        \
      acc[pt][(f0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[pt], acc[pt][(f0) + j], 0, 0, 0);   \
    _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)        \
      acc[pt][(f0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ax[pt], acc[pt][(f0) + j], 0, 0, 0);   \
    _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) _Pragma("unroll") for (int j = 0; j < FC; ++j)        \
      acc[pt][(f0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al2[pt], acc[pt][(f0) + j], 0, 0, 0);  \
  }
  // pixel fragments of the four tiles from LDS rows rbase + pt * rstep + l15 of patch image P
#define UP_AFRAGS(P, rbase, rstep)                                                                         \
  f16x8u ah[4], ax[4], al2[4];                                                                             \
  _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) {                                                       \
    int R_ = (rbase) + pt * (rstep) + l15;                                                                 \
    asm volatile("" : "+v"(R_));                                                                           \
    const float* ap_ = (P) + R_ * 32;                                                                      \
    const int sw_ = R_ & 7;                                                                                \
    ah[pt] = *reinterpret_cast<const f16x8u*>(ap_ + ((slot_hi ^ sw_) << 2));                               \
    const f16x8u al_ = *reinterpret_cast<const f16x8u*>(ap_ + ((slot_lo ^ sw_) << 2));                     \
    ax[pt] = ah[pt] * (_Float16)U_LO_INV;                                                                  \
    al2[pt] = al_ * (_Float16)U_LO_INV;                                                                    \
  }

  // ---- phase A: 3x3 taps over the skip patch -------------------------------------------------------------------------
  // tile y' of this wave, tap (r, s): halo row 2 y' + py + r, columns 2 l15 + px + s: parity (px + s) & 1, index
  // l15 + ((px + s) >> 1) -> LDS row (2 y' + py + r) * 34 + parity * 17 + index
  UP_ISSUE_A(0)
  UP_ISSUE_B(0, 0)
  UP_WAIT_ALL()
  int bs = 0;
  for (int cc = 0; cc < a.ncc1; ++cc) {
    if (cc > 0) {
      UP_ISSUE_A(cc)
      UP_WAIT_ALL()
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) {
        UP_ISSUE_B((tap + 1) * a.ncc_full + cc, bs ^ 1)
      } else if (cc + 1 < a.ncc1) {
        UP_ISSUE_B(cc + 1, bs ^ 1)
      }
      {
        const int r = tap / 3, s = tap % 3;
        const int rb = (py + r) * U_HW + ((px + s) & 1) * 17 + ((px + s) >> 1);
        UP_AFRAGS(As, rb, 2 * U_HW)
        const float* Bc = Bs + bs * B_STAGE;
#pragma unroll
        for (int f0 = 0; f0 < FT; f0 += FC) UP_MFMAS(Bc + f0 * 512, f0)
      }
      UP_WAIT_ALL()
      bs ^= 1;
    }
  }

  // ---- phase B: 2x2 taps over the source patch, class filters ---------------------------------------------------------
  // tile y' of this wave, tap (sy, sx): source patch row y' + sy + py, columns l15 + sx + px (patch origin = source
  // pixel (y0 / 2 - 1, x0 / 2 - 1))
  // Stage g = (cb, st): 2x2 tap st / SPT, filter block st % SPT.  While stage g is contracted, stages g + 1 .. g + 3 are
  // in flight; after it, `vmcnt(8)` (two stages of four DMA instructions may stay outstanding) retires stage g + 1.
  const int G = a.ncc2 * SPC;
  UP_ISSUE_S(0)
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (g < G) { UP_ISSUE_B2(g / SPT, g % SPT, 0, g) }            // (SPC >= 8 > 3: the first stages are all in block 0)
  }
  UP_WAIT_ALL()
  int slot = 0, gbase = 0;
  for (int cb = 0; cb < a.ncc2; ++cb, gbase += SPC) {
    if (cb > 0) {
      __builtin_amdgcn_s_barrier();            // everyone is past the last tap of the previous block: the patch is free
      asm volatile("" ::: "memory");
      UP_ISSUE_S(cb)
      UP_WAIT_ALL()                            // (also lands the class slices already in flight)
    }
#pragma unroll
    for (int tap2 = 0; tap2 < 4; ++tap2) {
      // the pixel fragments of a 2x2 tap serve all its filter stages
      const int sy = tap2 >> 1, sx = tap2 & 1;
      const int rb = (sy + py) * U_SW + sx + px;
      UP_AFRAGS(Ss, rb, U_SW)
#pragma unroll
      for (int fblk = 0; fblk < SPT; ++fblk) {
        const int st = tap2 * SPT + fblk;
        const int left = G - (gbase + st) - 1;                          // stages after this one
        if (left >= 3) {
          if (st + 3 < SPC) { UP_ISSUE_B2((st + 3) / SPT, (st + 3) % SPT, cb, (slot + 3) & 3) }
          else { UP_ISSUE_B2((st + 3 - SPC) / SPT, (st + 3 - SPC) % SPT, cb + 1, (slot + 3) & 3) }
        }
        {
          const float* Bc = B2 + slot * U_B2SLOT + wave * (32 * 32);          // this wave's class slice
          UP_MFMAS(Bc, fblk * 2)
        }
        UP_WAIT_OWN(left >= 3 ? 2 : left - 1)    // stage g + 1 has landed (this wave's own DMA: no barrier needed)
        slot = (slot + 1) & 3;
      }
    }
  }
#undef UP_ISSUE_A
#undef UP_ISSUE_S
#undef UP_ISSUE_B
#undef UP_ISSUE_B2
#undef UP_WAIT_ALL
#undef UP_WAIT_OWN
#undef UP_MFMAS
#undef UP_AFRAGS

  // ---- epilogue: lane = pixel (y0 + 2 y' + py, x0 + 2 l15 + px); filter tiles 2 u, 2 u + 1 give it channels
  // n0 + 32 u + 8 g4 .. + 7 (one S16 group, 32 bytes); the shift depends on the border class of the pixel
  float vmax = 0.f;
  const int xg = x0 + 2 * l15 + px;
  const int rx = xg == 0 ? 0 : (xg == d.width - 1 ? 2 : 1);
#pragma unroll
  for (int u = 0; u < FT / 2; ++u) {
    const int c0 = n0 + 32 * u + 8 * g4;
    float sc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sc[k] = 1.f;
    if (d.scale) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(d.scale + c0), s1 = *reinterpret_cast<const f32x4*>(d.scale + c0 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = s0[k], sc[4 + k] = s1[k];
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int yg = y0 + 2 * pt + py;
      const int ry = yg == 0 ? 0 : (yg == d.height - 1 ? 2 : 1);
      const float* shp = a.shift9 + (int64_t)(ry * 3 + rx) * d.n + c0;
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(shp), h1 = *reinterpret_cast<const f32x4*>(shp + 4);
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = acc[pt][2 * u + (k >> 2)][k & 3] * sc[k] + (k < 4 ? h0[k & 3] : h1[k & 3]);
        if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
        v[k] = t;
      }
      ammc_u4 hi, lo;
      ammc_s16_split8(v, hi, lo);
#pragma unroll
      for (int k = 0; k < 8; ++k) vmax = fmaxf(vmax, fabsf(v[k]));
      ammc_u4* yp = reinterpret_cast<ammc_u4*>(d.y + ((int64_t)b * d.y_bs + (int64_t)yg * d.y_rs + (int64_t)xg * d.y_ps) + c0);
      yp[0] = hi;
      yp[1] = lo;
    }
  }
  if (d.overflow_flag && !(vmax <= 65504.f)) atomicOr(d.overflow_flag, 1);
}

// composed filters: out[co][(tap2 * 4 + cls) * cin2 + ci], fp32 (the caller splits them into S16).
// One workgroup per output filter co: its 9 x c filter values of the up half sit in LDS; thread ci streams its own
// contiguous Wt[ci][:][2][2] (16 bytes per cu) once and accumulates all 16 (tap2, class) sums in double - every 3x3 tap
// (r, s) lands in exactly one tap2 per class, so a cu costs 36 multiply-adds.  (The first form, one thread per output
// element with strided reads of both operands, took 2.4 ms per decoder level.)
__global__ __launch_bounds__(256) void up_compose_kernel(const float* __restrict__ w3, const float* __restrict__ wt,
                                                         int n, int c, float* __restrict__ out) {
  extern __shared__ float a_s[];                       // [9][c]: W3[co][c + cu][r][s]
  const int cin2 = 2 * c;
  const int co = blockIdx.x;
  for (int i = threadIdx.x; i < 9 * c; i += 256) {
    const int rs = i / c, cu = i - rs * c;
    a_s[i] = w3[((int64_t)co * cin2 + c + cu) * 9 + rs];
  }
  __syncthreads();
  for (int ci = threadIdx.x; ci < cin2; ci += 256) {
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    const f32x4* wp = reinterpret_cast<const f32x4*>(wt + (int64_t)ci * c * 4);
    for (int cu = 0; cu < c; ++cu) {
      const f32x4 w4 = wp[cu];                          // [dy][dx]
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
              // up-row 2 y + py + r - 1: source row y + floor((py + r - 1) / 2) = y + sy + py - 1, dy = (py + r - 1) & 1
              const int ur = py + r - 1, uc = px + s - 1;
              const int sy = (ur < 0 ? -1 : (ur >> 1)) - (py - 1), sx = (uc < 0 ? -1 : (uc >> 1)) - (px - 1);
              const int dy = ur & 1, dx = uc & 1;
              const int q = (sy * 2 + sx) * 4 + py * 2 + px;
              acc[q] += (double)a_s[(r * 3 + s) * c + cu] * (double)w4[dy * 2 + dx];
            }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((int64_t)co * 16 + q) * cin2 + ci] = (float)acc[q];
  }
}

// shift9[ry][rx][co] = shift[co] + scale[co] * sum over the taps inside the image of W3[co][c + cu][r][s] * bt[cu]
__global__ __launch_bounds__(256) void up_shift9_kernel(const float* __restrict__ w3, const float* __restrict__ bt,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int n, int c, float* __restrict__ out) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= 9 * n) return;
  const int co = gid % n, cls = gid / n;
  const int ry = cls / 3, rx = cls % 3;
  double sum = 0.0;
  for (int r = 0; r < 3; ++r) {
    if ((ry == 0 && r == 0) || (ry == 2 && r == 2)) continue;
    for (int s = 0; s < 3; ++s) {
      if ((rx == 0 && s == 0) || (rx == 2 && s == 2)) continue;
      const float* w3p = w3 + (((int64_t)co * 2 * c + c) * 3 + r) * 3 + s;
      for (int cu = 0; cu < c; ++cu) sum += (double)w3p[(int64_t)cu * 9] * (double)bt[cu];
    }
  }
  const double sc = scale ? (double)scale[co] : 1.0, sh = shift ? (double)shift[co] : 0.0;
  out[gid] = (float)(sh + sc * sum);
}

template <int TN>
static int launch_up(const UpArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(U_SSTAGE + U_B2RING * U_B2SLOT) * sizeof(float);
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  auto kern = conv_up_s16_kernel<TN>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  UpArgs b = a;
  b.n_tiles = a.d.n / (32 * TN);
  const int grid = a.d.batch * a.tiles_y * a.tiles_x * b.n_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(U_NT), lds, stream, b);
  return ammc_launch_status();
}

}  // namespace ammc_s16
using namespace ammc_s16;

extern "C" int ammc_pack_up_conv_f32(const float* w3_oihw, const float* wt_iohw, const float* bt, const float* scale,
                                     const float* shift, int32_t n, int32_t c, float* w2_out, float* shift9_out,
                                     void* stream) {
  if (!w3_oihw || !wt_iohw || !bt || !w2_out || !shift9_out || n <= 0 || c <= 0) return AMMC_EINVAL;
  if (((uintptr_t)wt_iohw & 15) || (size_t)9 * c * sizeof(float) > 48 * 1024) return AMMC_EINVAL;
  hipLaunchKernelGGL(up_compose_kernel, dim3(n), dim3(256), (size_t)9 * c * sizeof(float), (hipStream_t)stream, w3_oihw,
                     wt_iohw, n, c, w2_out);
  hipLaunchKernelGGL(up_shift9_kernel, dim3((9 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3_oihw, bt, scale,
                     shift, n, c, shift9_out);
  return ammc_launch_status();
}

extern "C" int ammc_conv_up_s16(const AmmcConvDesc* desc, const float* up_x, int64_t up_bs, int64_t up_rs, int64_t up_ps,
                                int32_t up_cin, const float* up_w, const float* shift9, void* stream) {
  if (!desc || !desc->x || !desc->w || !desc->y || !up_x || !up_w || !shift9) return AMMC_EINVAL;
  const AmmcConvDesc& d = *desc;
  if (d.batch <= 0 || d.height <= 0 || d.width <= 0) return AMMC_EINVAL;
  if (d.ntaps != 9 || d.up != 1 || d.x_step > 1 || d.y_f32 || d.res || d.pool_y || d.n_store || d.y_cs > 1) return AMMC_EUNSUP;
  if (d.cin <= 0 || d.cin % 32 || up_cin != 2 * d.cin) return AMMC_EUNSUP;      // skip half c, source 2 c channels
  if (d.width % U_TW || d.height % U_TH) return AMMC_EUNSUP;
  if (d.n != 64 && d.n % 128) return AMMC_EUNSUP;
  if (((uintptr_t)d.x | (uintptr_t)d.w | (uintptr_t)up_x | (uintptr_t)up_w | (uintptr_t)shift9) & 15) return AMMC_EINVAL;
  if (((uintptr_t)d.y & 31) || ((d.y_bs | d.y_rs | d.y_ps) & 7)) return AMMC_EINVAL;
  if ((d.x_bs | d.x_rs | d.x_ps | up_bs | up_rs | up_ps) & 7) return AMMC_EINVAL;
  const int64_t patch = (int64_t)(U_TH + 1) * d.x_rs + (int64_t)(U_TW + 1) * d.x_ps;
  const int64_t spatch = (int64_t)(U_SH - 1) * up_rs + (int64_t)(U_SW - 1) * up_ps;
  const int64_t ymax = (int64_t)d.batch * d.y_bs;
  if (patch >= (1LL << 30) || spatch >= (1LL << 30) || ymax >= (1LL << 31)) return AMMC_EUNSUP;
  UpArgs a;
  a.d = d;
  a.x2 = up_x;
  a.x2_bs = up_bs, a.x2_rs = up_rs, a.x2_ps = up_ps;
  a.w2 = up_w;
  a.shift9 = shift9;
  a.cin2 = up_cin;
  a.ncc1 = d.cin / 32;
  a.ncc2 = up_cin / 32;
  a.ncc_full = 2 * d.cin / 32;                   // the packed 3x3 filter covers both halves of the concatenation
  a.kpad = 9 * 2 * d.cin;
  a.kpad2 = 16 * up_cin;
  a.tiles_x = d.width / U_TW;
  a.tiles_y = d.height / U_TH;
  a.n_tiles = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d.n == 64 ? launch_up<2>(a, s) : launch_up<4>(a, s);
}
